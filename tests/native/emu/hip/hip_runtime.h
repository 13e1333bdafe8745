// stands in for <hip/hip_runtime.h> in the CPU emulator build (tests/native/emu_build.py)
#pragma once
#include "../simt_emu.hpp"
