// The mixed-kind decoder launch of ragged batches (tdec_mix_kernel, see the end of tdec.hip's device part): a translation unit of its own that
// compiles tdec.hip's decoder BODIES - the pair-mapped one of tdec_pair.inc, the 8-window one, the unwindowed one - into one kernel. Kept apart
// so that the kernels of tdec.hip come out exactly as they do without it: with the mixed kernel in the same unit the inliner takes another
// order through the shared device functions and tdec_pair_kernel lost 1.5 % of the headline line (profiles/r04/ab_mix_kernel.txt).
#define TDEC_MIX_TU 1
#include "tdec.hip"
