set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$R"
OUT=gpurun_out/profiles_r03
mkdir -p $OUT
PM="env DCC_AUTOTUNE=0 python bench.py --steps 1 --warmup 1 --launches-per-step 4 --no-c3 --no-cpu-baseline --place-tries 0"
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- $PM > /dev/null 2> $OUT/pmc_$C.err
  python tools/pmc_summary.py $OUT/pmc_$C 150 > $OUT/pmc_$C.txt 2>&1
  rm -rf $OUT/pmc_$C $OUT/pmc_$C.err
done
cat $OUT/pmc_WRITE_SIZE.txt $OUT/pmc_FETCH_SIZE.txt
