#!/usr/bin/env python3
"""Writes the preset input files of the drop-in contract into ./inputs.

The toolkit's inputs are three ';'-separated text files: ``inputs/namelist`` (role -> variable name and units in the data
file), ``inputs/box_limits`` (fixed framework) and a track file (moving framework).  The presets below are the schema filled in
for the data sources the reference documents (ERA5 in its three download flavours, MPAS-A, NCEP-R1, NCEP-R2); a run copies
one of them to ``inputs/namelist`` exactly as the reference's tests do (tests/test_R2_fixed.py:11-12 there).
The names are facts about those data sets' files, so the tables agree with the reference's presets."""
import os

ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "inputs")

ROLES = [("Air Temperature", "air_temperature"), ("Geopotential", "geopotential"), ("Geopotential Height", "geopotential_height"),
         ("Specific Humidity", "specific_humidity"), ("Omega Velocity", "omega"), ("Eastward Wind Component", "eastward_wind"),
         ("Northward Wind Component", "northward_wind")]
COORDS = ["Longitude", "Latitude", "Time", "Vertical Level"]

# data source -> ({role: (variable, units[, standard_name override])}, (lon, lat, time, level) coordinate names)
SI = {"T": "K", "Z": "m**2/s**2", "W": "Pa/s", "U": "m/s", "V": "m/s"}
PRESETS = {
    "ERA5": ({"Air Temperature": ("T", "K"), "Geopotential": ("Z", "m**2/s**2"), "Omega Velocity": ("W", "Pa/s"),
              "Eastward Wind Component": ("U", "m/s"), "Northward Wind Component": ("V", "m/s")}, ("longitude", "latitude", "time", "level")),
    "ERA5-cdsapi": ({"Air Temperature": ("t", "K"), "Geopotential": ("z", "m**2/s**2"), "Omega Velocity": ("w", "Pa/s"),
                     "Eastward Wind Component": ("u", "m/s"), "Northward Wind Component": ("v", "m/s")},
                    ("longitude", "latitude", "valid_time", "pressure_level")),
    "ERA5-copernicus": ({"Air Temperature": ("t", "K"), "Geopotential": ("z", "m**2/s**2"), "Specific Humidity": ("q", "kg/kg"),
                         "Omega Velocity": ("w", "Pa/s"), "Eastward Wind Component": ("u", "m/s"), "Northward Wind Component": ("v", "m/s")},
                        ("longitude", "latitude", "time", "level")),
    "ERA5-copernicus-new": ({"Air Temperature": ("t", "K"), "Geopotential": ("z", "m**2/s**2"), "Omega Velocity": ("w", "Pa/s"),
                             "Eastward Wind Component": ("u", "m/s"), "Northward Wind Component": ("v", "m/s")},
                            ("longitude", "latitude", "valid_time", "pressure_level")),
    "MPAS-A": ({"Air Temperature": ("tempk", "K"), "Geopotential": ("geop", "meter ** 2 / second ** 2"), "Omega Velocity": ("omega", "Pa/s"),
                "Eastward Wind Component": ("uwnd", "m/s"), "Northward Wind Component": ("vwnd", "m/s")}, ("longitude", "latitude", "Time", "level")),
    "NCEP-R1": ({"Air Temperature": ("air", "K"), "Geopotential Height": ("hgt", "m"), "Omega Velocity": ("omega", "Pa/s"),
                 "Eastward Wind Component": ("uwnd", "m/s", "u"), "Northward Wind Component": ("vwnd", "m/s", "v")}, ("lon", "lat", "time", "level")),
    "NCEP-R2": ({"Air Temperature": ("TMP_2_ISBL", "K"), "Geopotential Height": ("HGT_2_ISBL", "m"), "Omega Velocity": ("V_VEL_2_ISBL", "Pa/s"),
                 "Eastward Wind Component": ("U_GRD_2_ISBL", "m/s"), "Northward Wind Component": ("V_GRD_2_ISBL", "m/s")},
                ("lon_2", "lat_2", "initial_time0_hours", "lv_ISBL3")),
}

BOXES = {"box_limits": (-60, -30, -42.5, -17.5), "box_limits_Reg1": (-60, -30, -42.5, -17.5), "box_limits-testcase": (-53, -44, -31, -24)}

# stationary 15 x 15 degree box over the test samples' domain: five steps, 6-hourly (NCEP-R2 sample) and hourly (ERA5 sample)
TRACKS = {
    "track_testdata_NCEP-R2": [(f"2005-08-08-{h:02d}00" if h < 24 else "2005-08-09-0000", -22.5, -45) for h in (0, 6, 12, 18, 24)],
    "track_testdata_ERA5": [(f"2005-08-09-{h:02d}00", -22.5, -45) for h in range(5)],
}


def main():
    os.makedirs(ROOT, exist_ok=True)
    for name, (fields, coords) in PRESETS.items():
        lines = [";standard_name;Variable;Units"]
        for role, std in ROLES:
            if role in fields:
                var, units, *over = fields[role]
                lines.append(f"{role};{over[0] if over else std};{var};{units}")
        lines += [f"{role};;{c}" for role, c in zip(COORDS, coords)]
        with open(os.path.join(ROOT, f"namelist_{name}"), "w") as f:
            f.write("\n".join(lines) + "\n")
    for name, (w, e, s, n) in BOXES.items():
        with open(os.path.join(ROOT, name), "w") as f:
            f.write(f"min_lon;{w}\nmax_lon;{e}\nmin_lat;{s}\nmax_lat;{n}\n")
    for name, rows in TRACKS.items():
        with open(os.path.join(ROOT, name), "w") as f:
            f.write("time;Lat;Lon\n" + "".join(f"{t};{la};{lo}\n" for t, la, lo in rows))
    with open(os.path.join(ROOT, ".gitignore"), "w") as f:
        f.write("# the active copies a run works with\nnamelist\ntrack\n")


if __name__ == "__main__":
    main()
