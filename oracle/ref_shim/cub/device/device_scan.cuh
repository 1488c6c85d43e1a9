/* TEST INFRASTRUCTURE (oracle/_ref), see ../../cuda_runtime.h. */
#pragma once
#include <hipcub/device/device_scan.hpp>
