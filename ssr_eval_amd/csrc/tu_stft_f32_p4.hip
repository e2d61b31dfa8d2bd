// libssrhip.so translation unit: STFT kernels, transform precision float, part 4 (see tu_stft.inc)
#define SSR_TU_T float
#define SSR_TU_PART 4
#include "tu_stft.inc"
