// The split-operand (BTSBOT_F16X2) instantiation of the stage-1 megakernel, in a translation unit of its own
// (see the note at the end of stage0b.hip).
#define STAGE1B_X2_TU 1
#include "stage1b.hip"
