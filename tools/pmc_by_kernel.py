"""Per-kernel averages of a rocprofv3 --pmc run (counter_collection.csv) for the rollout3 instances, per dispatch and per env-step
of a 256-step x 4096-env launch:  python tools/pmc_by_kernel.py <rocprof output dir>
e.g.  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace
      --output-format csv -d gpurun_out/polpmc -- python3 tools/policy_rollout_bench.py"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "rollout3_kernel" in k:
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, "dispatches", len(next(iter(d.values()))))
    for c, v in d.items():
        print("   %-24s %12.0f per dispatch  %8.1f per env-step (256 x 4096)" % (c, sum(v) / len(v), sum(v) / len(v) / (256 * 4096)))
