cd /root/repo
TUNE_MS=120 TUNE_ITERS=20 TUNE_TIMEOUT=3000 TUNE_DEST=gpurun_out/gemm_tunings_hifi.csv bash tools/tune_gemms.sh 2>&1 | tail -6
ab() {
  for f in dynamic-coverage-control_amd/config/gemm_tunings_gfx950.csv gpurun_out/gemm_tunings_hifi.csv dynamic-coverage-control_amd/config/gemm_tunings_gfx950.csv gpurun_out/gemm_tunings_hifi.csv; do
    DCC_TUNED_GEMMS_FILE=$PWD/$f python bench.py --mode mappo --iters 3 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['config']; print('$f'.split('/')[-1], '%.3f M update %.4f rollout %.4f'%(r['value']/1e6, c['update_s_per_iter'], c['rollout_s_per_iter']))"
  done
}
echo c3; ab
echo c4; ab --agents 16 --pois 256 --envs 1024
echo c5; ab --agents 32 --pois 1024 --envs 2048 --comm-force-scale 0.5 --r-comm 0.1
