mkdir -p gpurun_out
for r in 1 2; do for w in 3 40 200; do timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-check > gpurun_out/w_${w}_$r.log 2>&1; done; done
timeout -k 10 200 python bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-check > gpurun_out/w_3_k100.log 2>&1
