// conv_mfma_tail_dma_b.hip -- LDS-DMA tail kernels for image widths 32, 64 and 128.
#include "conv_mfma_tail_dma.h"

namespace glowhip {

int launch_tail_dma_wide(const TailConvArgs& a, int paired, hipStream_t s, int TP, int Y) {
    if (a.W == 128 && TP == 128) return launch_tail_dma<1, 2, 4, 1, 128>(a, paired, s, Y);
    if (a.W == 64 && TP == 128) return launch_tail_dma<1, 2, 4, 1, 64>(a, paired, s, Y);
    if (a.W == 64 && TP == 64) return launch_tail_dma<1, 1, 4, 1, 64>(a, paired, s, Y);
    if (a.W == 32 && TP == 128) return launch_tail_dma<1, 2, 4, 1, 32>(a, paired, s, Y);
    if (a.W == 32 && TP == 64) return launch_tail_dma<1, 1, 4, 1, 32>(a, paired, s, Y);
    if (a.W == 32 && TP == 32) return launch_tail_dma<1, 1, 2, 2, 32>(a, paired, s, Y);
    set_error("conv_mfma_tail: no LDS-DMA kernel for W=%d TP=%d", a.W, TP);
    return GLOWHIP_EINVAL;
}

}  // namespace glowhip
