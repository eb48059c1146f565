# Copy EVERYTHING a tools/collect_profiles.sh run collected for judging into profiles/ (build container, after gpurun):
#   bash tools/publish_profiles.sh TAG
cd "$(dirname "$0")/.."
TAG=${1:-r03_final}
P=gpurun_out/$TAG/publish
[ -f $P/${TAG}_manifest.json ] || { echo "no manifest under $P: the collection did not finish"; exit 1; }
cp $P/* profiles/
python - "$P/${TAG}_manifest.json" << 'PY'
import json, sys
sys.path.insert(0, ".")
from student_mechanism_design_amd.build import source_digest
import subprocess
m = json.load(open(sys.argv[1]))
m["git_head_at_publish"] = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
m["sources_unchanged_since_collection"] = m["library_build_id"] == source_digest()
json.dump(m, open("profiles/" + m["tag"] + "_manifest.json", "w"), indent=1)
print("published", len(m["files"]) + 1, "files; library build", m["library_build_id"], "| sources now:", source_digest(),
      "(same)" if m["library_build_id"] == source_digest() else "(DIFFERENT: the kernel changed since the collection)")
PY
