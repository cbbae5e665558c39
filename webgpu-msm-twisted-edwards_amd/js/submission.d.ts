// Type declaration matching the reference's export (submission/submission.ts:73-78).
export interface BigIntPoint { x: bigint; y: bigint; t: bigint; z: bigint; }
export interface U32ArrayPoint { x: Uint32Array; y: Uint32Array; t: Uint32Array; z: Uint32Array; }
export declare const compute_msm: (
  bufferPoints: BigIntPoint[] | U32ArrayPoint[] | Buffer,
  bufferScalars: bigint[] | Uint32Array[] | Buffer,
  log_result?: boolean,
  force_recompile?: boolean,
) => Promise<{ x: bigint; y: bigint }>;
// Not in the reference: the GPUs a call is sharded over (default: TE_MSM_DEVICES, else device 0).
export declare const setDevices: (ids: number[]) => void;
export declare const getDevices: () => number[];
// Not in the reference: resident bases.  setBases(points) binds that Buffer once (upload + conversion on every device);
// compute_msm(points, scalars) with the same Buffer object then moves the scalars only (full_benchmarks.ts:63-68,100-105 pass
// one point buffer to six calls per size).  setBases(null) unbinds.
export declare const setBases: (bufferPoints: Buffer | null) => void;
export declare const getStats: () => { submittedInEnter: number; submittedInExecute: number; loneRuns: number; boundJobs: number; maxInFlight: number };
