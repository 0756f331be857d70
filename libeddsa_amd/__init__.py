"""libeddsa_amd - Python host mirror of the MI355X Ed25519 / X25519 engine.

Thin ctypes binding over the C-ABI library ``libeddsa_amd.so`` (include/eddsa.h,
include/eddsa_amd.h; the test and measurement surface of include/eddsa_amd_debug.h is bound too).  Names, argument meaning and error behaviour mirror the reference's
public header (reference lib/eddsa.h:44-113); the ``*_batch`` functions are the batched
forms.  There is no CPU implementation behind any of them: importing works anywhere, but the
first call that needs the GPU raises :class:`EddsaAmdError` if the library or a gfx950
device is missing.
"""
from .api import (  # noqa: F401
    EddsaAmdError,
    DH,
    HOOKS_OFF,
    RLC_MIN_ITEMS_DEFAULT,
    STALLED,
    debug_withhold_handoff,
    debug_fail_hip_call,
    debug_fail_next_host_call,
    debug_halve,
    debug_hip_calls,
    debug_init,
    debug_layer,
    debug_teardown_errors,
    halve_rejected,
    host_array,
    host_free,
    set_host_threads,
    combiner_stats,
    ed25519_genpub,
    ed25519_genpub_batch,
    ed25519_sign,
    ed25519_sign_batch,
    ed25519_verify,
    ed25519_verify_batch,
    ed25519_verify_batch_multi,
    ed25519_verify_batch_rlc,
    ed25519_verify_batch_multi_dev,
    ed25519_sign_batch_multi,
    x25519_batch_multi,
    init_devices,
    device_count,
    secret_residue,
    ed25519_verify_records,
    eddsa_genpub,
    eddsa_pk_eddsa_to_dh,
    eddsa_sign,
    eddsa_sk_eddsa_to_dh,
    eddsa_verify,
    init,
    library,
    library_path,
    use_debug_library,
    pk_ed25519_to_x25519,
    pk_ed25519_to_x25519_batch,
    sk_ed25519_to_x25519,
    set_offcurve_mode,
    set_profiling,
    set_rlc_min_items,
    set_verify_algo,
    shutdown,
    sk_ed25519_to_x25519_batch,
    verify_phase_ms,
    x25519,
    x25519_base,
    x25519_base_batch,
    x25519_batch,
)
from .sharding import gather_bytes, shard_bounds  # noqa: F401

ED25519_KEY_LEN = 32
ED25519_SIG_LEN = 64
X25519_KEY_LEN = 32
