# where the wave-specialised split-bf16 kernel spends its time: SVA_SPLIT_DBG=1 idles the producer waves, 2 the consumer waves (timing only, results invalid)
for d in 0 1 2; do echo "== SVA_SPLIT_DBG=$d"; SVA_SPLIT_DBG=$d SVA_TUNE_TABLE=0 SVA_SPLIT_VARIANT=4 python tools/split_probe.py x 2>&1 | grep -v amdgpu | grep "(64, 1" | head -5; done
