// sf_setup.hpp -- initial data of the benchmark setup generated in device memory.
//
// Reference: initial_condition(), sailfish/solvers/cbdiso_2d.py:33-46 -- a Python double loop over every zone
// calling CircumbinaryDisk.primitive (sailfish/setups/circumbinary_disk.py:82-93), 2.7e8 interpreter calls at
// 16384^2 -- followed by the host-to-device copy of the array (cbdiso_2d.py:92-94).  The isothermal disk's
// formulas use only + * / sqrt max, all correctly rounded here as on the host (contraction off), so the
// device-made array is bit-identical to the loop's.  The gamma-law disk calls pow/exp, which no device
// library reproduces bit for bit: that branch stays on the host (sailfish_amd/setups.py).
#pragma once
#include "sf_device.hpp"

namespace sf {

// prim: (ni + 4, nj + 4, 3) with guards; rows are zones [i0, i0 + ni) of the mesh; interior zones only.
__global__ void disk_primitive_kernel(int ni, int nj, int i0, double x0, double y0, double dx, double dy,
                                      double softening2, double sigma, double *__restrict__ prim)
{
#pragma clang fp contract(off)
    const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (j >= nj) return;
    const double y = y0 + ((double)j + 0.5) * dy; // mesh.cell_coordinates, mesh.py:246-252
    for (long i = blockIdx.y; i < ni; i += gridDim.y) {
        const double x = x0 + ((double)(i0 + i) + 0.5) * dx;
        const double r2 = x * x + y * y;
        const double r = sqrt(r2);
        const double rs = sqrt(r2 + softening2);
        const double rc = r > 1e-12 ? r : 1e-12; // max(r, 1e-12)
        const double vk = sqrt(1.0 / rs);
        double *p = prim + ((i + NG) * (long)(nj + 2 * NG) + (j + NG)) * 3;
        p[0] = sigma;
        p[1] = vk * (-y / rc);
        p[2] = vk * (x / rc);
    }
}

} // namespace sf
