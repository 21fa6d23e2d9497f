R=$PWD; O=$R/gpurun_out/r3m; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/coarse -- python $R/bench.py --stream coarse --no-cpu-baseline > $O/bench_coarse.json 2> $O/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/coarse256 -- python $R/bench.py --stream coarse --frames 256 --no-cpu-baseline > $O/bench_coarse_t256.json 2>> $O/err.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
ls $O/coarse/runc
