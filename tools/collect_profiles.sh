# Collect the judged evidence for the default bench workload (run through gpurun):
#   kernel-trace stats, HBM traffic counters (separate FETCH_SIZE / WRITE_SIZE passes), SQ / LDS counters, the bench line.
# Output under gpurun_out/final/; copy into profiles/ afterwards (tools/collect_profiles.sh prints the copy commands).
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/final; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu --no-verify --steps 20 --warmup 5 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py --no-cpu --no-verify --steps 3 --warmup 1 --repeats 1 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py --no-cpu --no-verify --steps 3 --warmup 1 --repeats 1 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
bash tools/collect_pmc.sh tools/pmc_sq.txt final_sq || exit 1
bash tools/collect_pmc.sh tools/pmc_lds.txt final_lds || exit 1
python tools/pmc_traffic.py $O $O/pmc_traffic.json 256 || exit 1
cp $O/pmc_traffic.json profiles/pmc_traffic.json   # bench.py reads this for roofline.traffic (labelled "stored")
python bench.py --step-api > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cat $O/bench.json
