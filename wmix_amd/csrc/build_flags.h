// build_flags.h -- what this build of libwmix_amd.so is (round-4 VERDICT "weak" 8: a library made with a timing experiment's switch loaded,
// passed build()'s symbol check and produced wrong audio, and nothing said which build was loaded).
//
//  * Switches that change RESULTS for the sake of a timing experiment (WMX_AEC_EXP, WMX_AEC_EXP_BARRIERS, WMX_NS_EXP) do not compile
//    unless the build says what it is: -DWMX_TIMING_ONLY_BUILD.
//  * Every developer flag of the build (the Makefile's EXTRA, e.g. -DWMX_AEC_WAVES=5, -ffp-contract=fast) is recorded and returned by
//    wmx_build_info(); the product build returns "default".  wmix_amd/_lib.py refuses to load anything else unless
//    WMIX_AMD_ALLOW_VARIANT_BUILD=1, and bench.py prints it on its line.
#pragma once

#if (defined(WMX_AEC_EXP) || defined(WMX_AEC_EXP_BARRIERS) || defined(WMX_NS_EXP) || defined(WMX_AEC_EXP_NOFARFFT)) && !defined(WMX_TIMING_ONLY_BUILD)
#error "WMX_AEC_EXP / WMX_AEC_EXP_BARRIERS / WMX_NS_EXP give WRONG results (timing experiments): add -DWMX_TIMING_ONLY_BUILD to say so"
#endif

#ifndef WMX_BUILD_EXTRA
#define WMX_BUILD_EXTRA ""
#endif
