// tests/link/ref_cvector_main.cpp -- TEST INFRASTRUCTURE.  Entry point for the reference's tests/test_cvector.cpp (compiled from
// where it lies, never copied) linked against this repository's libstenos.so: stands in for the driver CMake generates
// (tests/CMakeLists.txt:8-52), as ref_tests_main.cpp does for the round-trip test.
int test_cvector(int, char*[]);
int main(int argc, char* argv[]) { return test_cvector(argc, argv); }
