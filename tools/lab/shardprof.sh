R=$PWD; OUT=$R/gpurun_out/r3shard; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace -d $OUT/kt -o c -- python $R/bench.py --no-cpu-baseline --no-secondary --min-seconds 0 --emulate-shard 0/8 --steps 400 --warmup 10 > $OUT/kt.log 2>&1
DB=$(find $OUT/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $OUT/shard8_kernel_stats.csv
python $R/tools/rocpd_stats.py $DB --busy 0.3 0.6 > $OUT/shard8_busy.txt
python $R/tools/rocpd_stats.py $DB --timeline k_build_fragments -20 > $OUT/shard8_timeline.csv
rm -rf $OUT/kt
tail -2 $OUT/kt.log
