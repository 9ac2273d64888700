// hiperror.h -- what the drop-in classes do when liborbhip reports an error.
//
// The reference's ORBextractor::operator() and ORBmatcher::Search* never throw (empty image: silent return,
// src/ORBextractor.cc:1048-1049; everything else is unconditional arithmetic) and their callers -- the Tracking,
// LocalMapping and LoopClosing threads -- have no try block.  An exception out of a drop-in would therefore be
// std::terminate of the whole SLAM process.  So, by default, a failed device call (HIP error, a documented size limit,
// no device) is turned into the result the caller already handles: no keypoints / released descriptors, 0 matches,
// untouched outputs.  The message is kept per thread (OrbHipLastError), counted (OrbHipErrorCount) and printed to
// stderr the first time each call site fails.  There is NO CPU fallback: a frame whose extraction failed is a lost frame
// ("too few matches" -> the reference's own LOST / relocalisation handling, src/Tracking.cc:1152-1168).
//
// Build with -DORBHIP_THROW to get std::runtime_error instead (the parity tests do: a silent empty result must not pass
// for a device that was never reached).
#ifndef ORBHIP_HIPERROR_H
#define ORBHIP_HIPERROR_H

namespace ORB_SLAM2
{
// message of the last failed drop-in call on the calling thread ("" if none failed yet)
const char *OrbHipLastError();
// number of failed drop-in calls in this process
unsigned long OrbHipErrorCount();

namespace hipdetail
{
// Records "who: msg"; throws under ORBHIP_THROW; always returns false so that call sites can `return Fail(...), 0`.
bool Fail(const char *who, const char *msg);
}
}  // namespace ORB_SLAM2
#endif
