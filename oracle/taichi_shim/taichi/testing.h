// oracle/taichi_shim (TEST INFRASTRUCTURE): forwards to the one-file stand-in for the un-vendored legacy taichi core.
#pragma once
#include <taichi/common/util.h>
