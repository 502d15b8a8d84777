// tests/link/ref_tests_shard_main.cpp -- TEST INFRASTRUCTURE.  The reference's own round-trip test
// (/root/reference/tests/tests_comp_decomp.cpp, compiled from where it lies -- it is included here as a translation
// unit, never copied) one (distribution, bytesoftype) cell at a time: its own driver walks "same", then "sorted", then
// "random" over bytesoftype 1..15 and needs hours to leave the first distribution, so a bounded run never saw the other
// two.  usage: ref_tests_shard <same|sorted|random> <bytesoftype 1..15>
#include "tests/tests_comp_decomp.cpp"

#include <stdlib.h>

template <size_t K>
static void one_cell(const char* distribution)
{
	TestDistribution<K, K + 1>::apply(distribution); // (the reference's own template: sizes, levels, threads, shrinking dst_size)
}
int main(int argc, char* argv[])
{
	if (argc < 3)
		return 2;
	const char* d = argv[1];
	switch (atoi(argv[2])) {
		case 1: one_cell<1>(d); break;
		case 2: one_cell<2>(d); break;
		case 3: one_cell<3>(d); break;
		case 4: one_cell<4>(d); break;
		case 5: one_cell<5>(d); break;
		case 6: one_cell<6>(d); break;
		case 7: one_cell<7>(d); break;
		case 8: one_cell<8>(d); break;
		case 9: one_cell<9>(d); break;
		case 10: one_cell<10>(d); break;
		case 11: one_cell<11>(d); break;
		case 12: one_cell<12>(d); break;
		case 13: one_cell<13>(d); break;
		case 14: one_cell<14>(d); break;
		case 15: one_cell<15>(d); break;
		default: return 2;
	}
	return 0;
}
