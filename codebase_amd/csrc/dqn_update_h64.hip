// hidden 64, LBF observation widths
#define MARL_PART_NAME lossgrad_part_h64
#define MARL_PART_SHAPES(X) X(12, 64, 6) X(15, 64, 6) X(18, 64, 6) X(21, 64, 6) X(24, 64, 6) X(27, 64, 6) X(39, 64, 6)
#include "dqn_update_part.h"
