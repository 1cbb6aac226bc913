"""Where a run's wall clock goes (tools/bench_cli.py): with ``LEC_PHASES=<file>`` in the environment the command line stamps the
end of each phase -- interpreter + imports, library load + HIP initialisation, opening the file (container / HDF5 chunk-index
parse) and planning, ingest + kernels (+ gather), track diagnostics, CSV writes -- and writes them as JSON when it ends.  Without the
variable ``mark`` does nothing.  Synchronisation points are the ones the program has anyway (the results reach the host before
the CSVs are written); nothing is added to the run."""
import json
import os
import time

_PATH = os.environ.get("LEC_PHASES")
_MARKS = []


def enabled() -> bool:
    return bool(_PATH)


def mark(name: str) -> None:
    if _PATH:
        _MARKS.append((name, time.time()))


def dump(extra=None) -> None:
    if not _PATH:
        return
    t_start = None
    try:
        import psutil
        t_start = psutil.Process().create_time()
    except Exception:
        pass
    out = {"process_start": t_start, "marks": _MARKS, "extra": extra or {}}
    with open(_PATH, "w") as fh:
        json.dump(out, fh)
