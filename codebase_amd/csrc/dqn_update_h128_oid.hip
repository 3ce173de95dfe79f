// hidden 128, the additional widths of env.observe_id (D + P)
#define MARL_PART_NAME lossgrad_part_h128_oid
#define MARL_PART_SHAPES(X) X(14, 128, 6) X(17, 128, 6) X(25, 128, 6) X(31, 128, 6) X(47, 128, 6)
#include "dqn_update_part.h"
