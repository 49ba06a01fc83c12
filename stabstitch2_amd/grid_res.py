# mesh resolution of the reference (Full_model_inference/Codes/grid_res.py:2-3): 7 x 9 = 63 control points
GRID_H = 6
GRID_W = 8
