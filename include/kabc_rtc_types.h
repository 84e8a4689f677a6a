/* kabc_rtc_types.h -- the fixed-width integer names under hipRTC (run-time compilation of user
 * DeviceCosts, kabc_compile_cost_plugin): hipRTC compiles device code only and has no system
 * headers; its own definitions live in a private namespace. */
#ifndef KABC_RTC_TYPES_H
#define KABC_RTC_TYPES_H
#ifdef __HIPCC_RTC__
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
typedef unsigned long uintptr_t;
#endif
#endif
