// options.h -- the switches of the library that are not part of gcwt_params.
//
// The product library reads exactly two environment variables, both budgets that change nothing
// in the results: GHOSTCWT_BATCH_BYTES (workspace a batch of segments may take) and
// GHOSTCWT_STAGE_FLOATS (size of the pinned staging tiles of host results).  Everything that
// selects another kernel instantiation or planning layout is an explicit, named option set through
// gcwt_debug_set_option (include/ghostcwt_debug.h; tests) and read when a plan is created.
// Options that change accuracy or exist for measurements only (kMeasureOnly below) are accepted by
// libghostcwt_measure.so alone, which also takes any option from the environment as
// GHOSTCWT_<NAME> (tools/).
#pragma once

namespace gcwt {

// value of option `name` (lower case) or dflt when it is not set
long long option_or(const char* name, long long dflt);
bool option_is_set(const char* name);
// 0 ok, -1 unknown name, -2 measure build only; clear = back to the default
int option_set(const char* name, long long value, bool clear);

}  // namespace gcwt
