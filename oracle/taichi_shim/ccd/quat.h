// oracle/taichi_shim (TEST INFRASTRUCTURE): see ccd/ccd.h
#pragma once
#include <ccd/ccd.h>
