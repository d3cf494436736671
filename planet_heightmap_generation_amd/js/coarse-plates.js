// Drop-in for the projection half of the reference's js/coarse-plates.js.
//
//   projectCoarsePlates(mesh, r_xyz, coarseMesh, coarse_xyz, coarse_r_plate, seed, numPlates) -> Int32Array
//       (js/coarse-plates.js:51-117) — HIP kernel, one thread per hi-res cell; plate ids bit-exact.
//
// generateCoarsePlates (plate seeds, motion and ocean/land on the fixed 20 000-cell mesh, js/coarse-plates.js:19-42)
// is host logic of the reference and is not replaced: pass its outputs straight in.
import addon, { planetFor } from './native.js';

export function projectCoarsePlates(mesh, r_xyz, coarseMesh, coarse_xyz, coarse_r_plate, seed, numPlates) {
    const planet = planetFor(mesh, r_xyz, null);
    return addon.projectCoarsePlates(planet, coarseMesh.adjOffset, coarseMesh.adjList, coarse_xyz, coarse_r_plate, seed,
                                     numPlates == null ? null : numPlates);
}
