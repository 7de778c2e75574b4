// Exact genotype caller kernels (reference calling/exact.py).  Filled in below.
#pragma once
#include "denovo_kernel.hpp"
