// conv_mfma_tail_dma_a.hip -- LDS-DMA tail kernels for image widths 8 and 16 (the deep levels).
#include "conv_mfma_tail_dma.h"

namespace glowhip {

int launch_tail_dma_narrow(const TailConvArgs& a, int paired, hipStream_t s, int TP, int Y) {
#define GH_TAIL_DMA(w)                                                                \
    if (a.W == w) {                                                                   \
        if (TP == 128) return launch_tail_dma<1, 2, 4, 1, w>(a, paired, s, Y);        \
        if (TP == 64) return launch_tail_dma<1, 1, 4, 1, w>(a, paired, s, Y);         \
        if (TP == 32) return launch_tail_dma<1, 1, 2, 2, w>(a, paired, s, Y);         \
        if (TP == 16) return launch_tail_dma<1, 1, 1, 4, w>(a, paired, s, Y);         \
    }
    GH_TAIL_DMA(16) GH_TAIL_DMA(8)
#undef GH_TAIL_DMA
    set_error("conv_mfma_tail: no LDS-DMA kernel for W=%d TP=%d", a.W, TP);
    return GLOWHIP_EINVAL;
}

}  // namespace glowhip
