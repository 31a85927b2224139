"""TEST INFRASTRUCTURE: builds the serial stand-in of the device primitives (tests/hostsim) and tells the host mirror --
by an explicit call, there is no environment switch -- that THIS path may be loaded by this test process."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "hostsim", "_build", "libgrlbwt_sim.so")


def sim_library():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostsim")], stdout=subprocess.DEVNULL)
    from grlbwt_amd import engine
    engine._test_allow_standin(SIM)
    return SIM
