// gnnb_unity.hip -- every translation unit of the library as one (probe build only: `make probe`).
#include "k_prep.hip"
#include "k_aggregate.hip"
#include "k_gemm.hip"
#include "k_readout.hip"
#include "k_stack.hip"
#include "k_stack_zf.hip"
#include "k_misc.hip"
#include "gnnb_runtime.hip"
