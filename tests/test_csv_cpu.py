"""Format of `Nyxus.to_csv` against the reference's CSV writer (/root/reference/src/nyx/output_2_csv.cpp:420-755):
quoted header names (:470-484), quoted file names (:563), integer label / time (:573), values through "%g" (:430)."""
import numpy as np
import pandas as pd

from nyxus_amd.nyxus import Nyxus


def test_csv_format(tmp_path):
    df = pd.DataFrame({"intensity_image": ["p0_y1_r1_c0.ome.tif", "p0_y1_r1_c0.ome.tif"], "mask_image": ["seg.ome.tif", "seg.ome.tif"],
                       "ROI_label": np.array([1, 4000000000], np.uint32), "t_index": [0.0, 0.0],
                       "MEAN": [2.894736842105263, 1e-7], "GLCM_ASM_0": [0.137778, 123456789.0]})
    p = tmp_path / "out.csv"
    Nyxus.to_csv(df, str(p))
    lines = p.read_text().splitlines()
    assert lines[0] == '"intensity_image","mask_image","ROI_label","t_index","MEAN","GLCM_ASM_0"'
    assert lines[1] == '"p0_y1_r1_c0.ome.tif","seg.ome.tif",1,0,2.89474,0.137778'       # printf("%g"): six significant digits
    assert lines[2] == '"p0_y1_r1_c0.ome.tif","seg.ome.tif",4000000000,0,1e-07,1.23457e+08'
    assert len(lines) == 3
