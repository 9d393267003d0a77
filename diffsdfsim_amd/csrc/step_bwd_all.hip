// step_bwd.hip with every primitive SDF compiled in: provides launch_bwd_pre_all.  See the note at the top of step_bwd.hip.
#define DSS_ALL_SHAPES 1
#include "step_bwd.hip"
