// oracle/spgrid_ref_driver.cpp — TEST INFRASTRUCTURE.  Prints the REFERENCE's own sort key for grid coordinates:
// SparseMask::Linear_Offset(i, j, k) >> SparseMask::data_bits, with SparseMask = SPGrid_Mask<5, 5, 3> as
// src/mpm.h:73-75 instantiates it for GridState<3> (32 B, src/mpm_fwd.h:69-119) — the quantity
// sort_particles_and_populate_grid sorts by (src/mpm.cpp:785-790).  Compiled against the vendored header-only
// SPGrid where it lies under /root/reference (`make -C oracle ref_spgrid` -> oracle/_ref/spgrid_keys).
// stdin: "i j k" per line; stdout: "i j k key block_xbits block_ybits block_zbits" per line.
#include <SPGrid/Core/SPGrid_Mask.h>

#include <cstdint>
#include <cstdio>

int main() {
  using Mask = SPGrid::SPGrid_Mask<5, 5, 3>;
  int i, j, k;
  while (std::scanf("%d %d %d", &i, &j, &k) == 3)
    std::printf("%d %d %d %llu %d %d %d\n", i, j, k, (unsigned long long)(Mask::Linear_Offset(i, j, k) >> Mask::data_bits),
                (int)Mask::block_xbits, (int)Mask::block_ybits, (int)Mask::block_zbits);
  return 0;
}
