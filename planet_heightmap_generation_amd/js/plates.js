// Drop-in for smoothAndReconnectPlates of the reference's js/plates.js (:241-348): majority-vote smoothing of plate
// boundaries, then re-attachment of fragments cut off from their plate's largest component.  Native host stage
// (the passes are order-defined and run in place); r_plate is mutated like in the reference, nothing is returned.
import addon from './native.js';

export function smoothAndReconnectPlates(mesh, r_plate, plateSeeds, numPasses) {
    if (!(r_plate instanceof Int32Array)) throw new TypeError('r_plate must be an Int32Array');
    const seeds = Int32Array.from(plateSeeds);          // Set or Array, in iteration order
    addon.smoothAndReconnectPlates(mesh.numRegions, mesh.adjOffset, mesh.adjList, r_plate, seeds, numPasses);
}
