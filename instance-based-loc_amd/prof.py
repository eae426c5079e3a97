"""In-process kernel timing for bench.py's roofline line (HIP events on the launch stream)."""
_STATE = {"roofline": None}


def reset():
    _STATE["roofline"] = None


def roofline():
    return _STATE["roofline"]
