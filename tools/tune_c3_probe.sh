set -x
cd /root/repo
export PYTORCH_TUNABLEOP_FILENAME=/root/repo/gpurun_out/tunableop_c3.csv
# 1. default
python bench.py --mode mappo --iters 2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['config']; print('DEFAULT', r['value'], c['update_s_per_iter'], c['rollout_s_per_iter'])"
# 2. tune (eager rollout so that nothing is tuned under stream capture)
t0=$(date +%s)
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=5 timeout 1500 python bench.py --mode mappo --iters 1 --no-graph > gpurun_out/tune_run.log 2>&1
echo "tuning run took $(( $(date +%s) - t0 )) s"; tail -2 gpurun_out/tune_run.log | cut -c1-300
ls -la gpurun_out/tunableop_c3*.csv; wc -l gpurun_out/tunableop_c3*.csv
# 3. tuned, lookup only
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 python bench.py --mode mappo --iters 2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['config']; print('TUNED', r['value'], c['update_s_per_iter'], c['rollout_s_per_iter'])"
python bench.py --mode mappo --iters 2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['config']; print('DEFAULT again', r['value'], c['update_s_per_iter'], c['rollout_s_per_iter'])"
