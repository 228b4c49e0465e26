# per-level kernel time of the fused vocoder levels under variants:  tools/ab_voc.sh
run() {  # tag, streams, steps, env...
  local TAG=$1 B=$2 K=$3; shift 3
  env "$@" bash tools/prof_steady.sh $TAG $B $K > /dev/null 2>&1
  echo "== $TAG B=$B $@"; grep -E "voc_level" gpurun_out/${TAG}_steady_kernel_stats.csv | sed -E 's/.*voc_level_kernel<([0-9]+)>[^,]*,/C=\1 /' | cut -c1-60
}
run w1_b1 1 60 SVA_VOC_WLDS=1
run w1_b1_tr32 1 60 SVA_VOC_WLDS=1 SVA_VOC_TR=32
run w0_b1 1 60 SVA_VOC_WLDS=0
run w1_b4 4 40 SVA_VOC_WLDS=1
run g_b4 4 40 SVA_VOC_FUSED=0
grep -E "M,N|,16,|,32," gpurun_out/x 2>/dev/null | head -0
timeout 300 python -m pytest tests -m gpu -x -q -k "vocoder or stream_vs_reference" 2>&1 | tail -2
