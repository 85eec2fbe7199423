// Internal: the kernels share the public C-ABI structs (include/grl_hip.h).
#pragma once
#include "../../include/grl_hip.h"
