// lds_dma.hpp -- global -> LDS without registers (gfx950: global_load_lds_dwordx4), shared by gmmmap_screen.hpp and
// traj_solve_blk.hpp
#pragma once

namespace vcmi {
// one 1 KB wave instruction of LDS-DMA: uniform global address, uniform LDS byte address, the lane's 16-byte offset
__device__ __forceinline__ void dma_1k(const char *ga, unsigned la, unsigned lane_off) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(ga), "s"(la) : "memory", "m0");
}
}  // namespace vcmi
