// rejit_amd/csrc/dense_streams_select.hip -- the SELECT instantiations of dense_streams.hip (the bit-stream dense kernel with the
// reference's left-most-longest selection made inside it: patterns whose candidates can overlap, StreamPlan::select) as a
// translation unit of their own, so that the two halves of the template's 96 kernels compile side by side.
#define RJ_DENSE_SELECT_TU 1
#include "dense_streams.hip"
