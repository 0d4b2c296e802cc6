// shadows <hip/hip_runtime.h> in the CPU emulation build (tests/cpu/hipemu/Makefile puts this directory first on the include path)
#pragma once
#include "../hipemu.h"
