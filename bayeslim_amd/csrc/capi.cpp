// capi.cpp -- version / diagnostics entry points of librime_hip.so
#include "rime_common.h"

namespace rime { char g_last_error[256] = ""; }

extern "C" const char* rime_version(void) { return "rime_hip 0.1.0 gfx950"; }
extern "C" const char* rime_last_error(void) { return rime::g_last_error; }
