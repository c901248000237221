"""yoloret_amd: the YOLO-ReT detection forward path on MI355X (see README.md)."""
import os

# DetectionPipeline(depth > 1) keeps several steps in flight on their own HIP streams, next to the streams of an overlapped
# all-gather and of a host feeder.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in the
# order of their first launch; streams that share a queue wait for each other.  Eight queues keep them apart (bench.py sets
# the same; effective only if the package is imported before the process makes its first HIP call).
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
