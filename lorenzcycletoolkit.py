#!/usr/bin/env python3
"""Lorenz Energy Cycle (LEC) program -- MI355X-native engine behind the command line of
daniloceano/LorenzCycleToolkit (reference lorenzcycletoolkit.py:50-265).

Same flags, same ``inputs/namelist`` / ``inputs/box_limits`` / track files, same
``./LEC_Results/<infile>_<method>/`` tree and CSV schema; the numerics run as HIP kernels on the GPU
(``lorenzcycletoolkit_amd``).  Out of scope: plots (-p), the interactive
domain chooser (-c), CDS-API downloads (--cdsapi).
"""
import argparse
import logging
import os
import sys
import time

import pandas as pd

from lorenzcycletoolkit_amd.dataset import prepare_data
from lorenzcycletoolkit_amd.frameworks import lec_fixed, lec_moving


def create_arg_parser():
    """Reference lorenzcycletoolkit.py:50-129 (identical flags and defaults)."""
    parser = argparse.ArgumentParser(description="Lorenz Energy Cycle (LEC) program.")
    parser.add_argument("infile", help="Input .nc file with temperature, geopotential/geopotential height, and wind components data.")
    parser.add_argument("-r", "--residuals", action="store_true", help="Compute the Dissipation and Generation terms as residuals.")
    group = parser.add_mutually_exclusive_group(required=True)
    group.add_argument("-f", "--fixed", action="store_true", help="Compute the energetics for a fixed domain specified by the 'box_limits' file.")
    group.add_argument("-t", "--track", action="store_true", help="Define the domain using a track file.")
    group.add_argument("-c", "--choose", action="store_true", help="Interactively select the domain for each time step.")
    parser.add_argument("-z", "--zeta", action="store_true", help="Use the vorticity from the track file instead of computing it at 850 hPa.")
    parser.add_argument("-m", "--mpas", action="store_true", help="Specify this flag if working with MPAS-A data processed with MPAS-BR routines.")
    parser.add_argument("-p", "--plots", action="store_true", help="Generate plots.")
    parser.add_argument("-v", "--verbosity", action="store_true", help="Logger level set to debug mode.")
    parser.add_argument("--cdsapi", action="store_true", help="Use CDS API for downloading data (experimental).")
    parser.add_argument("--time-resolution", type=int, default=3, help="Temporal resolution in hours for CDS API data download (default: 3).")
    parser.add_argument("--trackfile", type=str, default="inputs/track", help="Specify a custom track file. Default is 'inputs/track'.")
    parser.add_argument("--box_limits", type=str, default="inputs/box_limits", help="Specify a custom box limits file. Default is 'inputs/box_limits'.")
    parser.add_argument("--device-ingest", action="store_true", help="(MI355X engine, with -f) stream the file bytes to the GPU and decode / sort / crop them there instead of preparing the data on the host.")
    parser.add_argument("-o", "--outname", type=str, help="Specify an output name for the results.")
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


def initialize_logging(results_subdirectory, args):
    """Reference src/utils/tools.py:32-73: logger "lorenzcycletoolkit", file log.<stem> + console."""
    level = logging.DEBUG if args.verbosity else logging.INFO
    logger = logging.getLogger("lorenzcycletoolkit")
    logger.setLevel(level)
    for h in list(logger.handlers):
        logger.removeHandler(h)
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
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
    args = create_arg_parser().parse_args(argv)
    method = "fixed" if args.fixed else ("track" if args.track else "choose")
    results_subdirectory, figures_directory, results_subdirectory_vertical_levels = setup_results_directory(args, method)
    app_logger = initialize_logging(results_subdirectory, args)
    app_logger.info("Starting LEC analysis")
    app_logger.info(f"Command line arguments: {args}")
    try:
        if args.device_ingest:
            from lorenzcycletoolkit_amd.ingest import prepare_streamed
            data = prepare_streamed(args, "inputs/namelist", app_logger)
        else:
            data = prepare_data(args, "inputs/namelist", app_logger)
        try:
            run_lec_analysis(data, args, results_subdirectory, figures_directory, results_subdirectory_vertical_levels, app_logger)
        finally:
            if args.device_ingest:
                data.raw.close()
    except Exception:
        app_logger.exception("LEC analysis failed")
        raise


if __name__ == "__main__":
    main(sys.argv[1:])
