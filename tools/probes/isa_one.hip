// One rollout3 instance alone, for a quick look at its ISA (tools/isa_only.sh); not part of the library.
#include "../../student_mechanism_design_amd/csrc/sy_rollout3.hpp"
#ifndef SY_ISA_NR
#define SY_ISA_NR 4
#endif
#ifndef SY_ISA_REC
#define SY_ISA_REC true
#endif
#ifndef SY_ISA_PT
#define SY_ISA_PT 4
#endif
#ifndef SY_ISA_POL
#define SY_ISA_POL false
#endif
#ifndef SY_ISA_HS
#define SY_ISA_HS 2
#endif
template __global__ void sy::rollout3_kernel<SY_ISA_NR, SY_ISA_REC, SY_ISA_PT, SY_ISA_POL, SY_ISA_HS>(const sy::EngineParams, const int, const sy_rollout_buffers);
