// oracle/ref_sky_tables.cpp -- TEST / BUILD INFRASTRUCTURE (compiled against /root/reference, never by the product).
//
// Writes tungsten_amd/data/skydome_tables.bin: the numeric tables behind the reference's `skydome` primitive
// (primitives/Skydome.cpp:255-317), which the host of this repository restates the model over (tungsten_amd/csrc/host/SkyModel.cpp):
//   * the coefficient tables of the Hosek-Wilkie sky model for its eleven spectral bands, 320 .. 720 nm -- `datasets[wl]` (2 albedos x
//     10 turbidities x 6 control points x 9 coefficients) and `datasetsRad[wl]` (2 x 10 x 6) of thirdparty/skylight/
//     ArHosekSkyModelData_Spectral.h, (c) 2012-2013 Lukas Hosek and Alexander Wilkie, 3-clause BSD licence (tungsten_amd/data/LICENSE.hosek)
//   * the CIE 1931 colour-matching functions at 1 nm from 360 to 830 nm the reference converts spectra with (math/Spectral.cpp:
//     CIE_X_entries / CIE_Y_entries / CIE_Z_entries, 471 entries each)
// Layout (little endian): char magic[8] = "TGSKY001"; double coeff[11][1080]; double rad[11][120]; float cie[3][471].
//
//   g++ -I/root/reference/src/core -I/root/reference/src/thirdparty oracle/ref_sky_tables.cpp oracle/_ref/libcore.a -o oracle/_ref/sky_tables
//   oracle/_ref/sky_tables tungsten_amd/data/skydome_tables.bin                      (oracle/Makefile.ref: target `sky-tables`)
#include <cstdio>
#include <cstring>
#include <skylight/ArHosekSkyModelData_Spectral.h>
#include "math/Spectral.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: sky_tables <out.bin>\n"); return 2; }
    std::FILE *f = std::fopen(argv[1], "wb");
    if (!f) { std::perror(argv[1]); return 1; }
    std::fwrite("TGSKY001", 1, 8, f);
    for (int wl = 0; wl < 11; ++wl) std::fwrite(datasets[wl], sizeof(double), 1080, f);
    for (int wl = 0; wl < 11; ++wl) std::fwrite(datasetsRad[wl], sizeof(double), 120, f);
    using namespace Tungsten::Spectral;
    std::fwrite(CIE_X_entries, sizeof(float), CIE_samples, f);
    std::fwrite(CIE_Y_entries, sizeof(float), CIE_samples, f);
    std::fwrite(CIE_Z_entries, sizeof(float), CIE_samples, f);
    std::fclose(f);
    return 0;
}
