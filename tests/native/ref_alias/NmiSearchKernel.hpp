// Test-only include alias: lets oracle/ref_search_kernel_driver.cpp -- written against the REFERENCE's class
// (Thirdparty/Localization/nmiSearchKernel.hpp) -- compile unchanged against this repository's class of the same name.
#pragma once
#include "nmi_search_kernel.hpp"
