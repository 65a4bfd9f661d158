// nmi_config.hpp -- the compile-time configuration surface of the reference, kept source-compatible.
//
// A host that already has Thirdparty/Localization/allProperties.hpp on its include path keeps using it
// (define NMI_HAVE_ALLPROPERTIES before including this file); otherwise the same macro names get the
// reference's default values (allProperties.hpp:27-50, kernel.cuh:22-28).  Every macro below only feeds a
// run-time default (nmi_params / nmi_properties); nothing in the library is compiled differently by them.
#pragma once
#ifdef NMI_HAVE_ALLPROPERTIES
#include "allProperties.hpp"
#endif

#ifndef nmi_prop_MAX_ITERATION_COUNT
#define nmi_prop_MAX_ITERATION_COUNT 4
#endif
#ifndef orb_prop_log
#define orb_prop_log false
#endif
#ifndef nmi_prop_RELOC_FREQUENCY
#define nmi_prop_RELOC_FREQUENCY 2
#endif
#ifndef nmi_prop_STEPFACTOR
#define nmi_prop_STEPFACTOR 0.5f
#endif
#ifndef nmi_prop_NUMBER_OF_THREADS
#define nmi_prop_NUMBER_OF_THREADS 12
#endif
#ifndef nmi_prop_BG
#define nmi_prop_BG true
#endif
#ifndef nmi_prop_RENDER
#define nmi_prop_RENDER 1
#endif
#ifndef nmi_prop_MIN_KERNEL_ROTATION
#define nmi_prop_MIN_KERNEL_ROTATION 0.001
#endif
#ifndef nmi_prop_MIN_KERNEL_TRANSLATION
#define nmi_prop_MIN_KERNEL_TRANSLATION 0.005
#endif

// Thirdparty/CUDA_Functions/kernel.cuh:22-28
#ifndef ENMI
#define ENMI 0
#endif
#ifndef SUC
#define SUC 1
#endif
#ifndef MATCHING_NMI
#define MATCHING_NMI 0
#define MATCHING_HOG 1
#define MATCHING_CANNY 2
#define MATCHING_HOUGH 3
#endif
