#!/bin/bash
# Audit for latent s_waitcnt placement bugs (one was found in gemm_f16w.hip this round): the same workloads through the normal
# library and through a build with -mllvm -amdgpu-waitcnt-forcezero (every counter drained after every instruction) must give
# bit-identical codes and PCM.   on the GPU box:  bash tools/waitcnt_audit.sh     (the forcezero library is built HERE beforehand:
#   make -C streamvoiceanon_amd/csrc BUILD=build_fz LIB=../libsva_hip_fz.so EXTRA="-mllvm -amdgpu-waitcnt-forcezero=1")
mkdir -p gpurun_out/waitcnt
# (round 4 ran both arms with SVA_DEBUG=ar_batch=0: the forcezero build of ar_batch.hip raised HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION.  Round 5 root cause: that
#  compiler mode puts an s_waitcnt between s_getpc_b64 / s_add_u32 @rel32@lo+4 / s_addc_u32 @rel32@hi+12 of every device-function CALL sequence, whose +4 / +12
#  assume adjacency -- the s_swappc landed 4 bytes in front of nucleus_sample; the sampler is now always inlined (ar_device.h), no call sequences are left in
#  the library, and the audit covers the kernel that serves 5-24 streams by default)
for L in normal fz; do
  if [ $L = fz ]; then export SVA_LIB_PATH=$PWD/streamvoiceanon_amd/libsva_hip_fz.so; else unset SVA_LIB_PATH; fi
  timeout 1200 python tools/waitcnt_audit.py > gpurun_out/waitcnt/$L.txt 2> gpurun_out/waitcnt/$L.err
done
if cmp -s gpurun_out/waitcnt/normal.txt gpurun_out/waitcnt/fz.txt; then echo "IDENTICAL: $(wc -l < gpurun_out/waitcnt/normal.txt) checksums"; else echo "DIFFERENT"; diff gpurun_out/waitcnt/normal.txt gpurun_out/waitcnt/fz.txt | head -20; fi
