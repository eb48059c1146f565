# Collect the judged evidence for the default bench workload (run through gpurun):  bash tools/collect_profiles.sh [TAG]
#   kernel-trace stats, HBM traffic counters (separate FETCH_SIZE / WRITE_SIZE passes), SQ / LDS counters, the bench line.
# Everything that is to be judged lands under gpurun_out/TAG/publish/ with its final name, every file from ONE library
# build (manifest.json holds its sy_build_id; publish adds the git HEAD); afterwards, in the build container:
#   bash tools/publish_profiles.sh TAG        (copies publish/* into profiles/ in one step)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
TAG=${1:-r03_final}
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O/publish
P=$O/publish
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu --no-verify --no-config3 --steps 20 --warmup 5 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py --no-cpu --no-verify --no-config3 --steps 3 --warmup 1 --repeats 1 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py --no-cpu --no-verify --no-config3 --steps 3 --warmup 1 --repeats 1 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
bash tools/collect_pmc.sh tools/pmc_sq.txt ${TAG}_sq --no-config3 || exit 1
bash tools/collect_pmc.sh tools/pmc_lds.txt ${TAG}_lds --no-config3 || exit 1
python tools/pmc_traffic.py $O $P/pmc_traffic.json 256 || exit 1
cp $P/pmc_traffic.json profiles/pmc_traffic.json   # on the box: the bench run below reads it (same build -> same stamp)
python bench.py --step-api > $P/${TAG}_bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $P/${TAG}_kernel_stats.csv
cp $(ls $O/fetch/*/*_counter_collection.csv | head -1) $P/${TAG}_pmc_fetch.csv
cp $(ls $O/write/*/*_counter_collection.csv | head -1) $P/${TAG}_pmc_write.csv
for k in sq lds; do for n in 1 2; do cp gpurun_out/${TAG}_${k}_pass$n.csv $P/${TAG}_${k}_counters_pass$n.csv; done; done
python tools/sq_report.py $((256 * 4096)) $P/${TAG}_sq_counters_pass1.csv $P/${TAG}_sq_counters_pass2.csv $P/${TAG}_lds_counters_pass1.csv $P/${TAG}_lds_counters_pass2.csv > $P/${TAG}_sq_report.txt
python - "$P" "$TAG" << 'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
from student_mechanism_design_amd import _lib
P, tag = sys.argv[1], sys.argv[2]
json.dump({"tag": tag, "library_build_id": _lib.build_id(), "files": sorted(os.listdir(P))},
          open(os.path.join(P, tag + "_manifest.json"), "w"), indent=1)
PY
cat $P/${TAG}_bench.json
