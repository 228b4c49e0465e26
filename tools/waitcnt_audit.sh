#!/bin/bash
# Audit for latent s_waitcnt placement bugs (one was found in gemm_f16w.hip this round): the same workloads through the normal
# library and through a build with -mllvm -amdgpu-waitcnt-forcezero (every counter drained after every instruction) must give
# bit-identical codes and PCM.   on the GPU box:  bash tools/waitcnt_audit.sh     (the forcezero library is built HERE beforehand:
#   make -C streamvoiceanon_amd/csrc BUILD=build_fz LIB=../libsva_hip_fz.so EXTRA="-mllvm -amdgpu-waitcnt-forcezero=1")
mkdir -p gpurun_out/waitcnt
# (round 4: the batched persistent decode kernel, ar_batch.hip, does not survive the forcezero build -- the library built that way raises
#  HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION in it (tools/fz_probe.py; the normal build is unaffected) -- so both arms run the sizes it serves on the multi-launch decode)
export SVA_DEBUG=ar_batch=0
export AUDIT_SKIP_GROUP=1       # (ar_group.hip: the same fault; its parity with ar_decode.hip is a GPU test)
for L in normal fz; do
  if [ $L = fz ]; then export SVA_LIB_PATH=$PWD/streamvoiceanon_amd/libsva_hip_fz.so; else unset SVA_LIB_PATH; fi
  timeout 1200 python tools/waitcnt_audit.py > gpurun_out/waitcnt/$L.txt 2> gpurun_out/waitcnt/$L.err
done
if cmp -s gpurun_out/waitcnt/normal.txt gpurun_out/waitcnt/fz.txt; then echo "IDENTICAL: $(wc -l < gpurun_out/waitcnt/normal.txt) checksums"; else echo "DIFFERENT"; diff gpurun_out/waitcnt/normal.txt gpurun_out/waitcnt/fz.txt | head -20; fi
