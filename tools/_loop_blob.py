import sys, os, traceback
sys.path.insert(0, os.getcwd())
import tests.test_gpu_multi_device as T
bad = 0
for i in range(25):
    try:
        T.test_blob_records_one_gather_one_copy()
    except Exception as e:
        bad += 1
        print("iter", i, "FAILED:", repr(e)[:300]); traceback.print_exc(limit=3)
print("failures", bad, "of 25")
