// Recorded launch lists, replayed from ONE call ("host off the critical path without hipGraph").
//
// A train step of complex_yolov4.cfg is ~650 stream-ordered calls into this library whose arguments do not change from step to
// step (static storages, static plan, scalars on the device).  Issued from Python they cost 12.3 ms of interpreter + ctypes
// time per step; the hipGraph of the same step replays in 33 ms on ROCm 7.2.  Instead the operator layer
// RECORDS the calls of one eager pass -- entry point + argument values, stream-ordering calls (cy_event_record /
// cy_stream_wait_event) included -- and cy_run_plan re-issues them in C.  Nothing is captured by the runtime: a replay is
// exactly the sequence of launches the eager pass makes, so results are bit-identical and every kernel still runs eagerly.
// The reference has no counterpart (its step is eager PyTorch, src/train.py:205-235).
//
// Program layout (int64 words): [fn index, nargs, arg 0, ..., arg nargs-1] per call; pointers and integers as themselves,
// floats as the bits of a double.  plan_tramp.inc (generated from include/cyolo_hip.h by build.py) holds one trampoline per
// int-returning entry point that unpacks the words into the C signature.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <string.h>

#include "cyolo_hip.h"

namespace {
union Word {
    int64_t i;
    double d;
    void* p;
};
typedef int (*Tramp)(const Word*);
struct Entry {
    const char* name;
    Tramp fn;
    int nargs;
};
#include "plan_tramp.inc"
}  // namespace

extern "C" int cy_event_create(void** ev) {
    if (!ev) return CY_ERR_ARG;
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return CY_ERR_ARG;
    *ev = (void*)e;
    return 0;
}

extern "C" int cy_event_destroy(void* ev) { return ev && hipEventDestroy((hipEvent_t)ev) == hipSuccess ? 0 : CY_ERR_ARG; }

extern "C" int cy_event_record(void* ev, cy_stream_t s) {
    return ev && hipEventRecord((hipEvent_t)ev, (hipStream_t)s) == hipSuccess ? 0 : CY_ERR_ARG;
}

extern "C" int cy_stream_wait_event(cy_stream_t s, void* ev) {
    return ev && hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)ev, 0) == hipSuccess ? 0 : CY_ERR_ARG;
}

extern "C" int cy_plan_fn_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < (int)(sizeof(kEntries) / sizeof(kEntries[0])); ++i)
        if (!strcmp(kEntries[i].name, name)) return i;
    return -1;
}

extern "C" int cy_plan_fn_nargs(int fn) {
    return fn >= 0 && fn < (int)(sizeof(kEntries) / sizeof(kEntries[0])) ? kEntries[fn].nargs : -1;
}

extern "C" int cy_run_plan(const int64_t* prog, int64_t nwords, int32_t* failed_op) {
    const int nfn = (int)(sizeof(kEntries) / sizeof(kEntries[0]));
    int64_t at = 0;
    int32_t op = 0;
    while (at < nwords) {
        if (at + 2 > nwords) return CY_ERR_ARG;
        const int64_t fn = prog[at], na = prog[at + 1];
        if (fn < 0 || fn >= nfn || na != kEntries[fn].nargs || at + 2 + na > nwords) {
            if (failed_op) *failed_op = op;
            return CY_ERR_ARG;
        }
        const int rc = kEntries[fn].fn(reinterpret_cast<const Word*>(prog + at + 2));
        if (rc != 0) {
            if (failed_op) *failed_op = op;
            return rc;
        }
        at += 2 + na;
        ++op;
    }
    return 0;
}
