// sn_internal.h — entry points shared by the translation units of libstereonet_hip.so that are NOT part of the C ABI
// (include/stereonet_hip.h); hidden visibility: nothing outside the library can bind them.
#pragma once
#include "../../include/stereonet_hip.h"

// sn_create with the priority of the engine's pipeline streams chosen by the caller: 1 = the device's highest stream
// priority, 0 = default, -1 = as sn_create (SN_STREAM_PRIORITY decides; unset = default).  An explicit
// SN_STREAM_PRIORITY=0 / 1 overrides the argument.  Used by sn_mgpu_create, whose gather runs on streams of its own
// beside the engines' (stereonet_infer/src/stereonet_node.cpp:144: independent requests in flight, nothing shared).
extern "C" __attribute__((visibility("hidden"))) int sn_create_prio(const char* model_file, const sn_config* cfg,
                                                                     int stream_prio, sn_handle** out);
