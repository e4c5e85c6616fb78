"""Older A/B scripts set GHOSTCWT_<NAME> in the environment; since round 4 the library takes named options instead
(the measure build still reads the environment).  `apply_env_options()` forwards whatever the environment holds to
gcwt_debug_set_option, so that `GHOSTCWT_INTERP=0 python tools/level_table.py` keeps working with either library."""
import os


def apply_env_options():
    from ghost_amd.engine import set_option
    skip = {"LIB", "RDZV_DIR", "COMM", "RCCL_TIMEOUT", "ALLOW_SHARED_GPU", "BENCH_FAIL_RANK", "BATCH_BYTES", "STAGE_FLOATS"}
    for key, val in os.environ.items():
        if key.startswith("GHOSTCWT_") and key[9:] not in skip:
            try:
                set_option(key[9:].lower(), int(val) if val.lstrip("-").isdigit() else 1)
            except Exception as e:                     # measure-only option with the product library, or unknown
                print("option %s not applied: %s" % (key, str(e)[:80]))
