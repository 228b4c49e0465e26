mkdir -p gpurun_out/dt; rm -f gpurun_out/dt/*
for i in $(seq 24); do python tools/det_trace.py 2>/dev/null > gpurun_out/dt/run$i.txt; done
md5sum gpurun_out/dt/*.txt | awk '{print $1}' | sort | uniq -c
ref=$(md5sum gpurun_out/dt/*.txt | sort | awk '{print $1}' | uniq -c | sort -rn | head -1 | awk '{print $2}')
for f in gpurun_out/dt/*.txt; do if [ "$(md5sum $f | awk '{print $1}')" != "$ref" ]; then echo "--- $f"; good=$(md5sum gpurun_out/dt/*.txt | grep $ref | head -1 | awk '{print $2}'); diff $good $f | head -8 | cut -c1-400; fi; done
