"""Import-path shim: put `compat/` and `pose-estimation_amd/` on PYTHONPATH and the reference's
callers (`from openpose.opencv import ...`, `from openpose.native.python.skeleton import ...`) get the posehip
estimators instead of the cv2.dnn / pyopenpose ones.  Only the estimator modules are shimmed; the reference's
own host utilities (args.py, utils.py, utils_rs.py, tracking/) are not part of this repository."""
