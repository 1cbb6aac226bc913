#!/usr/bin/env python3
"""Lorenz Energy Cycle (LEC) program -- MI355X-native engine behind the command line of
daniloceano/LorenzCycleToolkit (reference lorenzcycletoolkit.py:50-265).

Same flags, same ``inputs/namelist`` / ``inputs/box_limits`` / track files, same
``./LEC_Results/<infile>_<method>/`` tree and CSV schema; the numerics run as HIP kernels on the GPU
(``lorenzcycletoolkit_amd``).  Out of scope: plots (-p), the interactive
domain chooser (-c), CDS-API downloads (--cdsapi).

Several GPUs of one node: ``python lorenzcycletoolkit.py <file> -r -f --gpus N`` (this process starts N rank processes, one per
GPU) or ``python -m torch.distributed.run --nproc-per-node N lorenzcycletoolkit.py <file> -r -f``.  The time steps are sharded
over the ranks (each reads, decodes and computes only its own block plus a one-step halo of T), rank 0 gathers the per-step
results over RCCL and writes the very same files as the one-GPU run.
"""
import argparse
import logging
import os
import sys
import time

import pandas as pd

from lorenzcycletoolkit_amd import phases
from lorenzcycletoolkit_amd.dataset import prepare_data
from lorenzcycletoolkit_amd.frameworks import lec_fixed, lec_moving

phases.mark("imports")          # interpreter start -> here: Python, pandas, torch and the package


def create_arg_parser():
    """The reference's command line (lorenzcycletoolkit.py:50-129 there): the same flags, defaults and mutual exclusions -- the
    flag table is the drop-in contract; the help texts are this program's own."""
    parser = argparse.ArgumentParser(description="Lorenz Energy Cycle of a limited area, computed on an AMD GPU (MI355X HIP engine).")
    parser.add_argument("infile", help="NetCDF file (classic or NetCDF-4) holding T, u, v, omega and geopotential (or geopotential height) "
                        "on isobaric levels; variable names come from inputs/namelist")
    parser.add_argument("-r", "--residuals", action="store_true", help="close the budgets with residual terms (RGz, RKz, RGe, RKe) instead of "
                        "friction-based dissipation; the only mode the reference completes")
    group = parser.add_mutually_exclusive_group(required=True)
    group.add_argument("-f", "--fixed", action="store_true", help="Eulerian framework: one box for the whole series, read from the box-limits file")
    group.add_argument("-t", "--track", action="store_true", help="semi-Lagrangian framework: one box per time step, centred on the track file's positions")
    group.add_argument("-c", "--choose", action="store_true", help="pick each time step's box on a map (needs a GUI: not available in this build)")
    parser.add_argument("-z", "--zeta", action="store_true", help="with -t: report the 850-hPa vorticity at the track position rather than the box extremum")
    parser.add_argument("-m", "--mpas", action="store_true", help="input comes from MPAS-A post-processed with MPAS-BR")
    parser.add_argument("-p", "--plots", action="store_true", help="accepted for compatibility; figures are made by the reference's plot scripts from the CSVs")
    parser.add_argument("-v", "--verbosity", action="store_true", help="log at DEBUG level")
    parser.add_argument("--cdsapi", action="store_true", help="download ERA5 through the CDS API first (needs network access: not available in this build)")
    parser.add_argument("--time-resolution", type=int, default=3, help="hours between downloaded analyses with --cdsapi (default: 3)")
    parser.add_argument("--trackfile", type=str, default="inputs/track", help="track file for -t (default: inputs/track)")
    parser.add_argument("--box_limits", type=str, default="inputs/box_limits", help="box-limits file for -f (default: inputs/box_limits)")
    parser.add_argument("--device-ingest", action="store_true", help="stream the file's bytes to the GPU in chunks and decode / sort / crop them "
                        "there, instead of preparing the whole data set on the host (same results, bit for bit); the same as --ingest device")
    parser.add_argument("--ingest", choices=["auto", "host", "device"], default="auto", help="where the data are prepared: 'host' decodes, sorts "
                        "and crops with NumPy and uploads the cubes; 'device' streams the file's bytes and does it on the GPU; 'auto' (default) "
                        "takes the device for deflated NetCDF-4 files whose chunks the GPU can inflate (the host inflates them ten times "
                        "slower) and for any file of 1 GiB or more that the streamed path can read, the host otherwise -- the output files "
                        "are the same, byte for byte")
    parser.add_argument("--inflate", choices=["auto", "host", "device"], default="auto", help="with --device-ingest and a chunked NetCDF-4 file: where "
                        "the (deflated) chunks are inflated -- on the GPU (lec_inflate; the default wherever the variables allow it) or on the host's threads")
    parser.add_argument("--vorticity-form", choices=["metpy_no_crs", "spherical"], default="metpy_no_crs", help="with -t: formulation of the 850-hPa "
                        "relative vorticity in the trackfile (default: what MetPy 1.6.2 evaluates for data without a CRS, as the reference passes them)")
    parser.add_argument("--gpus", type=int, default=1, help="shard the time steps over this many GPUs of the node (one process per GPU, "
                        "results gathered over RCCL; same output files).  Under torch.distributed.run the launcher's WORLD_SIZE counts")
    parser.add_argument("-o", "--outname", type=str, help="name of the results CSV (fixed framework)")
    return parser


def setup_results_directory(args, method):
    """Reference lorenzcycletoolkit.py:132-155."""
    results_subdirectory = os.path.join("./LEC_Results/", "".join(args.infile.split("/")[-1].split(".nc")) + "_" + method)
    results_subdirectory_vertical_levels = os.path.join(results_subdirectory, "results_vertical_levels")
    figures_directory = os.path.join(results_subdirectory, "Figures")
    os.makedirs(figures_directory, exist_ok=True)
    os.makedirs(results_subdirectory, exist_ok=True)
    os.makedirs(results_subdirectory_vertical_levels, exist_ok=True)
    return results_subdirectory, figures_directory, results_subdirectory_vertical_levels


def _leave_only_the_log(results_subdirectory):
    """A run that failed leaves no half-written results behind.  Like the reference, the results tree is created BEFORE the input is
    read (lorenzcycletoolkit.py:250-258 there), so a file the reader refuses, a bad namelist or a failure in mid-analysis would
    leave empty directories and header-only CSVs that look like results.  When THIS run created the tree, everything but the log --
    which carries the error -- is removed again; a tree that existed before (earlier results) is not touched."""
    import shutil
    for name in os.listdir(results_subdirectory):
        path = os.path.join(results_subdirectory, name)
        if name.startswith("log."):
            continue
        if os.path.isdir(path):
            shutil.rmtree(path, ignore_errors=True)
        else:
            try:
                os.remove(path)
            except OSError:
                pass


def initialize_logging(results_subdirectory, args):
    """Reference src/utils/tools.py:32-73: logger "lorenzcycletoolkit", file log.<stem> + console.  In a time-sharded run only
    rank 0 logs to the file; the other ranks report warnings and errors on the console."""
    level = logging.DEBUG if args.verbosity else logging.INFO
    logger = logging.getLogger("lorenzcycletoolkit")
    logger.setLevel(level)
    for h in list(logger.handlers):
        logger.removeHandler(h)
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    shard = getattr(args, "shard", None)
    if shard is not None and not shard.root:
        ch = logging.StreamHandler()
        ch.setFormatter(logging.Formatter(f"%(asctime)s - rank {shard.rank} - %(levelname)s - %(message)s"))
        ch.setLevel(logging.WARNING)
        logger.addHandler(ch)
        return logger
    stem = os.path.basename(args.infile).split(".nc")[0]
    fh = logging.FileHandler(os.path.join(results_subdirectory, f"log.{stem}"), mode="w")
    fh.setFormatter(fmt)
    ch = logging.StreamHandler()
    ch.setFormatter(fmt)
    logger.addHandler(fh)
    logger.addHandler(ch)
    return logger


def run_lec_analysis(data, args, results_subdirectory, figures_directory, results_subdirectory_vertical_levels, app_logger):
    """Reference lorenzcycletoolkit.py:158-200."""
    start_time = time.time()
    variable_list_df = pd.read_csv("inputs/namelist", sep=";", index_col=0, header=0)
    if args.fixed:
        lec_fixed(data, variable_list_df, results_subdirectory, results_subdirectory_vertical_levels, app_logger, args)
        app_logger.info("Analysis complete! Fixed framework ran in %.2f seconds" % (time.time() - start_time))
    if args.track or args.choose:
        # dT/dt over the (track-selected) time axis is formed on the device inside the engine
        lec_moving(data, variable_list_df, None, results_subdirectory, figures_directory,
                   results_subdirectory_vertical_levels, app_logger, args)
        app_logger.info("Analysis complete! Moving framework ran in %.2f seconds" % (time.time() - start_time))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = create_arg_parser().parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # this process only starts the ranks (before anything touches a GPU) and waits for them
        from lorenzcycletoolkit_amd.parallel import launch_local_ranks
        sys.exit(launch_local_ranks(__file__, argv, args.gpus))
    if env_world is not None and args.gpus > 1 and int(env_world) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} contradicts WORLD_SIZE={env_world}")
    from lorenzcycletoolkit_amd.parallel import shard_from_env
    args.shard = shard_from_env()                      # None: the ordinary one-process run
    method = "fixed" if args.fixed else ("track" if args.track else "choose")
    tree_is_new = not os.path.isdir(os.path.join("./LEC_Results/", "".join(args.infile.split("/")[-1].split(".nc")) + "_" + method))
    if args.shard is None or args.shard.root:
        results_subdirectory, figures_directory, results_subdirectory_vertical_levels = setup_results_directory(args, method)
    else:                                              # the paths only: rank 0 creates the tree and writes every file
        results_subdirectory = os.path.join("./LEC_Results/", "".join(args.infile.split("/")[-1].split(".nc")) + "_" + method)
        results_subdirectory_vertical_levels = os.path.join(results_subdirectory, "results_vertical_levels")
        figures_directory = os.path.join(results_subdirectory, "Figures")
    app_logger = initialize_logging(results_subdirectory, args)
    if phases.enabled():             # (measurement runs only: brings the library load and the HIP context forward so that they show as a phase)
        import torch
        from lorenzcycletoolkit_amd import _lib
        _lib.load()
        torch.zeros(1, device=(args.shard.device if args.shard is not None else os.environ.get("LEC_DEVICE", "cuda:0")))
        torch.cuda.synchronize()
        phases.mark("library_and_hip_init")
    app_logger.info("Starting LEC analysis")
    app_logger.info(f"Command line arguments: {args}")
    if args.shard is not None:
        app_logger.info(f"Time-sharded run: {args.shard.world} ranks (backend {args.shard.backend}), one GPU each; rank 0 writes the results")
    try:
        opened, auto_chose = None, False
        if args.ingest == "device":
            args.device_ingest = True
        elif args.ingest == "auto" and not args.device_ingest:
            from lorenzcycletoolkit_amd.ingest import prefers_device_ingest
            args.device_ingest, opened = prefers_device_ingest(args, "inputs/namelist", keep_open=True, app_logger=app_logger)
            auto_chose = args.device_ingest
            if args.device_ingest:
                app_logger.info("The input is a deflated NetCDF-4 file whose chunks the GPU can inflate, or a large file: streaming it to the "
                                "GPU (--ingest device); --ingest host prepares the data on the host instead (same results)")
        analyse = lambda d: run_lec_analysis(d, args, results_subdirectory, figures_directory, results_subdirectory_vertical_levels, app_logger)
        data = None
        if args.device_ingest:
            from lorenzcycletoolkit_amd.ingest import StreamedRefusal, prepare_streamed, refusals
            refused = None
            try:
                with refusals():
                    data = prepare_streamed(args, "inputs/namelist", app_logger, raw=opened)
                phases.mark("open_and_plan")
                try:
                    analyse(data)
                finally:
                    data.raw.close()
            except StreamedRefusal as e:
                # A streamed path that --ingest auto chose BY ITSELF must not fail a run the host preparation can do: the file is closed, the
                # reason logged, and the data are prepared on the host.  Only a REFUSAL of the streamed path counts (raised around
                # prepare_streamed / lec_streamed: before the engine has produced anything) -- an error later in the run (CSV writing,
                # plotting) is an error, not a reason to analyse everything a second time.  Asked for explicitly (--ingest device /
                # --device-ingest) the refusal stands.
                if not auto_chose or args.shard is not None:       # (ranks of a sharded run must not part ways)
                    raise e.__cause__ if e.__cause__ is not None else e
                refused = str(e)
            if refused is not None:
                # (outside the handler: the traceback -- and through its frames the failed attempt's device buffers -- is released first)
                if data is None and opened is not None:
                    opened.close()
                app_logger.warning(f"--ingest auto: the streamed path refused this input ({refused}); preparing the data on the host instead")
                args.device_ingest, data = False, None
                import gc
                import torch
                gc.collect()
                if torch.cuda.is_available():
                    torch.cuda.empty_cache()
        if not args.device_ingest:
            data = prepare_data(args, "inputs/namelist", app_logger)
            phases.mark("open_decode_and_prepare")
            analyse(data)
        if args.shard is not None:
            args.shard.barrier()                       # the ranks leave together, after rank 0 has written the files
    except Exception:
        app_logger.exception("LEC analysis failed")
        if tree_is_new and (args.shard is None or args.shard.root):
            _leave_only_the_log(results_subdirectory)
        raise
    finally:
        phases.mark("end")
        phases.dump({"argv": argv})
        if args.shard is not None:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


if __name__ == "__main__":
    status = main(sys.argv[1:])
    # The results are on disk and the logs flushed: leave without the interpreter's teardown.  A run that pinned / registered tens of
    # GB of host memory and holds a HIP context spends 1-2.5 s there (freeing pinned blocks one by one, unloading the runtime) -- a
    # quarter of the wall clock of a 96-step ERA5 file (profiles/r04_notes.md section 6); the operating system reclaims it all at once.
    # Everything must be flushed BEFORE this line: os._exit skips atexit handlers and buffered file objects (phases.dump and the CSV
    # writers close their files; logging is shut down here).  main()'s return value is the exit status -- a failure raises and never
    # gets here, so the interpreter's ordinary exit reports it.
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(int(status or 0))
