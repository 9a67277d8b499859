// conv_mfma.hip -- placeholder while the generic path is brought up: nothing is "supported" yet,
// so plans route every convolution through conv_direct.hip.
#include "conv_mfma.h"

namespace glowhip {
bool conv_mfma_wide_supported(int, int, int, int, int) { return false; }
size_t conv_mfma_wide_packed_bytes(int, int, int) { return 0; }
int conv_mfma_wide_pack(const float*, int, int, int, float*, hipStream_t) { return GLOWHIP_EINVAL; }
int launch_conv_mfma_wide(const float*, long, const float*, const float*, const float*, float*, int, int, int, int, int,
                          int, hipStream_t) { return GLOWHIP_EINVAL; }
bool conv_mfma_tail_supported(int, int, int, int) { return false; }
size_t conv_mfma_tail_packed_bytes(int, int) { return 0; }
int conv_mfma_tail_pack(const float*, int, int, int, float*, hipStream_t) { return GLOWHIP_EINVAL; }
int launch_conv_mfma_tail(const TailConvArgs&, hipStream_t) { return GLOWHIP_EINVAL; }
}  // namespace glowhip
