"""jetson_slam_amd - MI355X-native ORB front-end + stereo matcher for Jetson-SLAM's hot path.

Product code: csrc/ (HIP kernels + C ABI, built into libjsorb.so) and the Python host mirror of the
reference interface (orb.py).  The CPU oracle lives in /oracle and is never imported from here.
"""
