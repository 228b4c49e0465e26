for mt in 1 2 4; do for kw in 4 8 16; do echo "== MT=$mt KW=$kw"; SVA_SKINNY_MT=$mt SVA_SKINNY_KW=$kw python tools/gemm_sweep3.py; done; done
