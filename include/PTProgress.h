// Progress callback of the render call.  Same class name and virtual signature as the
// reference's PTProgress.h:3-7 so that a caller's CPTCallback subclass compiles unchanged.
#pragma once

class CPTCallback
{
public:
	virtual void UpdateProgress(long nProgress, long nTotal) = 0;
	virtual ~CPTCallback() {}
};
