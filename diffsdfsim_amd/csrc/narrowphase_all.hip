// narrowphase.hip with every primitive SDF compiled in (rounded box, brick, bowl as well): the variant
// launch_find_contacts hands over to when DssWorld.shape_rare is set.  See the note at the top of narrowphase.hip.
#define DSS_ALL_SHAPES 1
#include "narrowphase.hip"
