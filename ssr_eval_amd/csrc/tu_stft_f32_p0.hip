// libssrhip.so translation unit: STFT kernels, transform precision float, part 0 (see tu_stft.inc)
#define SSR_TU_T float
#define SSR_TU_PART 0
#include "tu_stft.inc"
