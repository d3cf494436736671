// Device versions of the three CSR sweeps that are module-private in the reference's climate stage — same names,
// argument order and return values (a fresh Float32Array), so temperature.js / precipitation.js can import them instead
// of defining them (INTEGRATION.md shows the two import lines):
//   diffuseOceanWarmth(mesh, r_oceanWarmth, r_isLand, r_plateContinentality, passes)            js/temperature.js:19-66
//   computeWindConvergence(mesh, r_xyz, r_wind3dX, r_wind3dY, r_wind3dZ)                        js/precipitation.js:18-52
//   advectMoisture(mesh, r_xyz, r_heightKm, r_isLand, r_windE, r_windN, r_wind3dX, r_wind3dY, r_wind3dZ,
//                  r_oceanWarmth, r_coastDistLand, maxHops, avgEdgeKm)                          js/precipitation.js:59-195
// The mesh is bound to a device planet on first use (r_xyz is needed for that; diffuseOceanWarmth has no r_xyz
// argument, so the mesh must already be bound: any earlier terrain-post / elevation call, or bindMesh()).
import addon, { planetFor } from './native.js';

export function diffuseOceanWarmth(mesh, r_oceanWarmth, r_isLand, r_plateContinentality, passes) {
    return addon.diffuseOceanWarmth(planetFor(mesh), r_oceanWarmth || null, r_isLand, r_plateContinentality || null, passes);
}
export function computeWindConvergence(mesh, r_xyz, r_wind3dX, r_wind3dY, r_wind3dZ) {
    return addon.computeWindConvergence(planetFor(mesh, r_xyz), r_wind3dX, r_wind3dY, r_wind3dZ);
}
export function advectMoisture(mesh, r_xyz, r_heightKm, r_isLand, r_windE, r_windN, r_wind3dX, r_wind3dY, r_wind3dZ,
    r_oceanWarmth, r_coastDistLand, maxHops, avgEdgeKm) {                 // avgEdgeKm: unused, as in the reference's body
    return addon.advectMoisture(planetFor(mesh, r_xyz), r_heightKm, r_isLand, r_windE, r_windN, r_wind3dX, r_wind3dY, r_wind3dZ,
        r_oceanWarmth || null, r_coastDistLand, maxHops);
}
