// k_stack_zf_head.hip -- the instantiations of k_gcn2_zf WITH the MLP-head tail (option zf_head: conv stack + pooling + head in one
// launch), as a translation unit of their own: see the note at the top of k_stack_zf.hip.
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
#define ZF_TU_HEAD 1
#include "k_stack_zf.hip"
