/* Development-only entry point of librdm_hip.so: present ONLY in libraries built with RDM_DEV_VARIANTS=1 (`RDM_DEV_VARIANTS=1 python -m
 * md_rdm_amd.build`); the shipped library neither declares nor exports it.  Tools under tools/ use it for in-process A/B timing of the
 * measured alternatives DESIGN.md cites. */
#ifndef RDM_DEV_H
#define RDM_DEV_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* 0 = the shipped configuration.  7 generic instead of halo 3x3, 8 no forward pipelining, 11 hardware block order, 13 128x96 wgrad tiles only,
 * 14 default-priority side stream, 16/20 256x48 tiles on 1x1 convs, 21 256-pixel halo tiles only, 23 full-size wgrad tiles at small M, ...
 * Results never depend on it beyond float rounding. */
void rdm_debug_variant(int32_t v);
#ifdef __cplusplus
}
#endif
#endif
