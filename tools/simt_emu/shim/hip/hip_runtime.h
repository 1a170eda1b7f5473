/* Emulator stand-in for <hip/hip_runtime.h> (TEST TOOLING ONLY, see ../../hip_emu.h).
 * The product sources include <hip/hip_runtime.h> unconditionally; the emulator build
 * puts this directory first on the include path. */
#include "../../hip_emu.h"
