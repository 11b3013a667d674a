"""CPU: the host-side manifold interface of the C++ facade (include/localization/filters/State.hpp, MtkWrap.hpp):
the assertions the reference's own tests hold (test/MsckfUnitTest.cpp:61, :62, :66, :71, :110, :113), the text
round trips (operator<< / operator>>, State.hpp:202-210, 298-306, 483-507, 636-646) and the reference's model functions
pasted unchanged (tests/cpp/manifold_identities.cpp) -- no GPU, no HIP library involved."""
import subprocess

import facade_build


def test_manifold_identities_and_text_io_of_the_cpp_facade():
    exe = facade_build.build("manifold_identities", link_hip=False, std="c++14")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "all manifold identities hold" in out.stdout


def test_reference_model_program_compiles_against_the_facade():
    """The GPU program that pastes the reference's model functions links against libslk_hip.so (run under -m gpu)."""
    import __graft_entry__ as ge
    ge.build()
    facade_build.build("reference_models", std="c++14")
    facade_build.build("facade_scenarios")
