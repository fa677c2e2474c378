"""Generates the golden fixtures under tests/golden/ from DATA files the reference's own tests hold
(EnergyPlus 9.6 outputs, /root/reference/tests/*/eplusout.csv). Run in the build container only; the
GPU box has no /root/reference and uses the committed .npz files.

    python tests/golden/make_fixtures.py

wall_<dir>.npz       first 7000 rows (5000 warm-up + 2000 compared, validate_wall_heat_transfer.rs:669-673)
                     of the columns march_model reads: 1 wind speed, 2 wind direction [deg], 3 incident solar,
                     8 outdoor dry bulb, 11 outside net thermal radiation gain, 12 zone mean air temperature.
convection_<dir>.npz every 7th row of the columns calc_convection reads (validate_convection.rs:33-90):
                     2, 4, 5, 6, 8, 9, 10, 12.
"""
import csv
import os

import numpy as np

REF = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))


def read_csv(d):
    rows = []
    with open(os.path.join(REF, d, "eplusout.csv")) as f:
        r = csv.reader(f)
        next(r)
        for row in r:
            rows.append([float(x) for x in row[1:13]])
    return np.array(rows)  # column j of the CSV is [:, j-1]


def main():
    # the fourteen wall series the reference registers: twelve constructions x radiation cases
    # (validate_wall_heat_transfer.rs:817-994) and the tilted / horizontal walls (:792-815, geometry in
    # tests/{tilted,horizontal}/back.spl)
    for d in [c + "_" + r for c in ("massive", "mixed", "nomass")
              for r in ("full", "no_ir_no_solar", "no_ir_yes_solar", "yes_ir_no_solar")] + ["tilted", "horizontal"]:
        a = read_csv(d)[:7000]
        np.savez_compressed(os.path.join(HERE, "wall_%s.npz" % d), wind_speed=a[:, 0], wind_dir_deg=a[:, 1],
                            solar=a[:, 2], t_out=a[:, 7], ir_gain=a[:, 10], zone_t=a[:, 11])
    for d in ("massive_full", "tilted", "horizontal"):
        a = read_csv(d)[::7]
        np.savez_compressed(os.path.join(HERE, "convection_%s.npz" % d), wind_dir_deg=a[:, 1], t_in_surf=a[:, 3],
                            t_out_surf=a[:, 4], hs_in=a[:, 5], t_out=a[:, 7], surf_wind=a[:, 8], hs_out=a[:, 9],
                            zone_t=a[:, 11])


if __name__ == "__main__":
    main()
