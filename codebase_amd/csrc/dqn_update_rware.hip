// the warehouse shapes (71 observations, 5 actions), hidden 64 and 128
#define MARL_PART_NAME lossgrad_part_rware
#define MARL_PART_SHAPES(X) X(71, 64, 5) X(71, 128, 5)
#include "dqn_update_part.h"
