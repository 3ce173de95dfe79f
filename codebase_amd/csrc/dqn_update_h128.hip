// hidden 128 (the reference's default width), LBF observation widths
#define MARL_PART_NAME lossgrad_part_h128
#define MARL_PART_SHAPES(X) X(12, 128, 6) X(15, 128, 6) X(18, 128, 6) X(21, 128, 6) X(24, 128, 6) X(27, 128, 6) X(39, 128, 6)
#include "dqn_update_part.h"
