/* TEST INFRASTRUCTURE (oracle/_ref), see ../../cuda_runtime.h. */
#pragma once
#include <hipcub/block/block_reduce.hpp>
namespace cub = hipcub;
