# ISA of ONE rollout3 instance (seconds): bash tools/isa_only.sh [extra -D flags] -> /tmp/isa/fast.s
# SY_PT / SY_HS / SY_POL (env) pick the instance (default: the headline <4,true,4,false,2>); tools/isa_hot.py reads the result.
mkdir -p /tmp/isa && cd "$(dirname "$0")/.."
POL=${SY_POL:-false}; PB=0; [ "$POL" = "true" ] && PB=1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -DSY_ISA_ONLY -DSY_ISA_PT=${SY_PT:-4} -DSY_ISA_HS=${SY_HS:-2} -DSY_ISA_POL=$POL "$@" \
  tools/probes/isa_one.hip -o /tmp/isa/fast_all.s 2>&1 | grep -i "error"
K=_ZN2sy15rollout3_kernelILi4ELb1ELi${SY_PT:-4}ELb${PB}ELi${SY_HS:-2}E
awk -v k="$K" 'index($0, k) == 1 && /:/ {on=1} on{print} index($0, ".amdhsa_kernel " k) {on=0}' /tmp/isa/fast_all.s \
  | grep -v "^\s*\.\(loc\|cfi\|file\)" | grep -v "^\s*;\s*\(APP\|NO_APP\)" | sed 's/^\s*; SYHOT/SYHOT/' | grep -v "^\s*;" | sed 's/\s*;.*$//' > /tmp/isa/fast.s
grep -E "vgpr_count|sgpr_count|spill|scratch" /tmp/isa/fast_all.s | grep -i "rollout3\|^\s*;" | head -8
wc -l /tmp/isa/fast.s
