// Library identification (checked by the Python loader and the symbol-export test).
#include "common.hpp"
extern "C" const char* sessd_version(void) { return "sessd_hip 0.1 gfx950"; }
