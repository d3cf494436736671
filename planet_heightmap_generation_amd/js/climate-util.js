// Drop-in for smoothField of the reference's js/climate-util.js (:5-25): `passes` Jacobi sweeps of
// (self + neighbours) / (1 + degree) over the CSR mesh, rewriting `field` (Float32Array) in place; returns undefined.
// The mesh must already be bound to a device planet (any earlier terrain-post / elevation call, or bindMesh()).
// makeItczLookup and percentile are small host utilities of the same reference module and are not replaced.
import addon, { planetFor } from './native.js';

export function smoothField(mesh, field, passes) {
    if (!(field instanceof Float32Array)) throw new TypeError('field must be a Float32Array');
    addon.smoothField(planetFor(mesh), field, passes);
}
