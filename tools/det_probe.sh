# how often do two fresh processes disagree?  digests of tests/_determinism_worker.py under a few configurations:  tools/det_probe.sh N "ENV=.." ...
N=${1:-10}; shift
for cfg in "$@"; do
  echo "== $cfg"
  for i in $(seq $N); do env $cfg python tests/_determinism_worker.py 2>/dev/null | grep DIGEST | cut -c1-30,72-95; done | sort | uniq -c
done
