// fused collectors on the warehouse with 8 agents (every layout)
#define MARL_RW_COLLECT_BIG(X) X(8, 512)
#define MARL_RW_PART_IDQN rware_idqn_collect_p8
#define MARL_RW_PART_AC rware_ac_collect_p8
#define MARL_RW_PART_NEXT_IDQN (set_error("rware_idqn_collect: no fused collector for %d agents on a %dx%d grid, hidden %d", q.n_agents, q.rows, q.cols, s->hidden), -1)
#define MARL_RW_PART_NEXT_AC (set_error("rware_ac_collect: no fused collector for %d agents on a %dx%d grid, hidden %d", q.n_agents, q.rows, q.cols, s->hidden), -1)
#include "rware_collect_part.h"
