# failure rate of the in-process determinism loop under configuration variants:  tools/det_loop_variants.sh R "ENV=.." ...
R=${1:-150}; shift
for cfg in "$@"; do
  echo "== $cfg: $(env $cfg DET_R=$R timeout 1200 python tools/det_loop.py 2>&1 | grep -v amdgpu | tail -1)"
done
