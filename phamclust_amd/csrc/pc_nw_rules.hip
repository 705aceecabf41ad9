// pc_nw_rules.hip -- explicit instantiations of the systolic kernel for two tie rules (PC_RULE_A, PC_RULE_B; default 2, 3).
// build.py compiles this file three times (-DPC_RULE_A=2 -DPC_RULE_B=3, 4/5, 6/7) next to pc_nw.hip, which holds rules 0, 1.
#include "pc_nw_systolic.h"
#ifndef PC_RULE_A
#define PC_RULE_A 2
#define PC_RULE_B 3
#endif
#define PC_INST1(W) template int pc_systolic_launch<W, PC_RULE_A, false> PC_SYSTOLIC_SIG; template int pc_systolic_launch<W, PC_RULE_B, false> PC_SYSTOLIC_SIG;
#define PC_INST_TIER(T) template int pc_tier_launch<T, PC_RULE_A, false> PC_TIER_SIG; template int pc_tier_launch<T, PC_RULE_B, false> PC_TIER_SIG; \
                        template int pc_tier_launch<T, PC_RULE_A, true> PC_TIER_SIG; template int pc_tier_launch<T, PC_RULE_B, true> PC_TIER_SIG;
PC_FOR_TIER(PC_INST_TIER)
PC_FOR_W1(PC_INST1)
// the strip-mined kernel (column genes beyond 64 x W columns): the three wide variants, and W = 24 with the profile cell (percent-positives)
#define PC_INST_STRIP(W, INC) template int pc_strip_launch<W, PC_RULE_A, INC> PC_STRIP_SIG; template int pc_strip_launch<W, PC_RULE_B, INC> PC_STRIP_SIG;
PC_INST_STRIP(32, false) PC_INST_STRIP(48, false) PC_INST_STRIP(64, false) PC_INST_STRIP(24, true) PC_INST_STRIP(8, false) PC_INST_STRIP(12, false)
// ... and its pipelined form (the passes of one alignment over the workgroup's waves), narrow passes only
template int pc_strip_launch<PC_STRIP_W_ONE_ROW, PC_RULE_A, false, true> PC_STRIP_SIG; template int pc_strip_launch<PC_STRIP_W_ONE_ROW, PC_RULE_B, false, true> PC_STRIP_SIG;
