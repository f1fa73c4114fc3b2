"""Tools only: library options from the environment variable HN_OPTIONS="name=value,name=value" (names: hermnet_amd/_lib.py:
OPTIONS) and host-code switches from HN_SWITCHES="boundary_mode=1,..." (hermnet_amd/switches.py) -- so that a shell A/B loop can
pick a kernel form without the PACKAGE reading any tuning knob from the environment."""
import os


def apply_option_env():
    from hermnet_amd import _lib, switches
    for item in filter(None, os.environ.get("HN_OPTIONS", "").split(",")):
        k, v = item.split("=")
        _lib.set_option(k.strip(), int(v))
    for item in filter(None, os.environ.get("HN_SWITCHES", "").split(",")):
        k, v = item.split("=")
        cur = getattr(switches, k.strip())
        setattr(switches, k.strip(), (v.strip() not in ("0", "False", "false")) if isinstance(cur, bool) else int(v))
