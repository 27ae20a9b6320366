cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06t
run() { # env..., -- args
  timeout 600 env "$@" python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r.get('avg_launch_us'), r['in_flight']['forward_us'], r['in_flight']['backward_us'], d['verified']['bit_exact'])"
}
for r in 1 2; do
for q in 4 3 5 6; do for s in 8 12; do
ARGS="--streams $s" ; echo "-- GPU_MAX_HW_QUEUES=$q, $s streams"; run GPU_MAX_HW_QUEUES=$q
done; done
done 2>&1 | tee gpurun_out/r06t/hwq2.txt
