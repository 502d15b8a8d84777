// tests/link/ref_tests_main.cpp -- TEST INFRASTRUCTURE.  Entry point for the reference's own round-trip test
// (/root/reference/tests/tests_comp_decomp.cpp, compiled from where it lies, never copied) linked against this
// repository's libstenos.so instead of the reference library: the reference builds its tests into one executable
// through CMake's create_test_sourcelist (tests/CMakeLists.txt:8-52); this file stands in for that generated driver.
int tests_comp_decomp(int, char*[]);
int main(int argc, char* argv[]) { return tests_comp_decomp(argc, argv); }
