cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  python bench.py --headline-only --no-cpu-baseline --steps 20 --warmup 5 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|step_ms_hip_events\": [0-9.]*" | tr '\n' ' '; echo " default"
  HSA_ENABLE_INTERRUPT=0 python bench.py --headline-only --no-cpu-baseline --steps 20 --warmup 5 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|step_ms_hip_events\": [0-9.]*" | tr '\n' ' '; echo " HSA_ENABLE_INTERRUPT=0"
done | tee gpurun_out/r06_w_sync_latency.txt
