"""HRT_TUNE (csrc/host/tune.c): the library's developer / test switches live in ONE environment variable,
"key=value,key=value"; later keys win.  `tuned(env, key=value, ...)` returns a copy of `env` with the keys
appended to whatever HRT_TUNE it already carries (tests/conftest.py sets rxt_min_rays=0 for the whole suite);
upper-case names are ordinary (supported) environment variables and are passed through."""
import os


def tuned(env=None, **keys):
    env = dict(os.environ if env is None else env)
    parts = [env["HRT_TUNE"]] if env.get("HRT_TUNE") else []
    for k, v in keys.items():
        if k.isupper():
            env[k] = str(v)
        else:
            parts.append("%s=%s" % (k, v))
    if parts:
        env["HRT_TUNE"] = ",".join(parts)
    return env


def without(env, *names):
    """`env` without the given tune keys (and without HRT_TUNE altogether if nothing is left)"""
    env = dict(env)
    keep = [p for p in env.get("HRT_TUNE", "").split(",") if p and p.split("=")[0] not in names]
    if keep:
        env["HRT_TUNE"] = ",".join(keep)
    else:
        env.pop("HRT_TUNE", None)
    return env
