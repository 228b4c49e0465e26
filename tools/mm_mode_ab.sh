#!/bin/bash
# frames/s against sva_config.mm_mode / voc_dtype at batch scale:  SIZES="16 32 64" CFGS="0,0 1,0 1,1" tools/mm_mode_ab.sh
for cfg in ${CFGS:-0,0 1,0 1,1}; do
  mm=${cfg%,*}; vd=${cfg#*,}
  echo "== mm_mode $mm voc_dtype $vd"
  BS="${SIZES:-16 32 64}" MM_MODE=$mm VOC_DTYPE=$vd bash tools/streams_curve.sh 2>&1 | cut -c1-230
done
