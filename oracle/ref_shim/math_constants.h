/* TEST INFRASTRUCTURE (oracle/_ref), see cuda_runtime.h in this directory. */
#pragma once
#include <math.h>
#define CUDART_NAN_F __builtin_nanf("")
#define CUDART_INF_F __builtin_inff()
#define CUDART_PI_F 3.141592654f
