// hidden 64, the additional widths of env.observe_id (D + P)
#define MARL_PART_NAME lossgrad_part_h64_oid
#define MARL_PART_SHAPES(X) X(14, 64, 6) X(17, 64, 6) X(25, 64, 6) X(31, 64, 6) X(47, 64, 6)
#include "dqn_update_part.h"
